#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <cstdlib>
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(int argc,char**argv){
  const char* dir=argv[1]; int T=atoi(argv[2]); size_t N=(size_t)atoll(argv[3])<<20; int mode=atoi(argv[4]);
  std::vector<char> src(64<<20, 'x');
  char path[256]; snprintf(path,256,"%s/wb.bin",dir);
  int fd=open(path,O_RDWR|O_CREAT|O_TRUNC,0644);
  double t0=now();
  if(mode==0){ // single thread write
    for(size_t o=0;o<N;o+=src.size()) (void)!pwrite(fd,src.data(),src.size(),o);
  } else if(mode==1){ // parallel pwrite 1MiB pieces
    std::vector<std::thread> th; size_t piece=1<<20; size_t np=N/piece;
    for(int t=0;t<T;t++) th.emplace_back([&,t]{ for(size_t i=t;i<np;i+=T) (void)!pwrite(fd,src.data()+(i%64)*piece,piece,i*piece);});
    for(auto&x:th)x.join();
  } else if (mode==2){ // mmap parallel
    (void)!ftruncate(fd,N); char* m=(char*)mmap(0,N,PROT_READ|PROT_WRITE,MAP_SHARED,fd,0);
    std::vector<std::thread> th; size_t piece=1<<20; size_t np=N/piece;
    for(int t=0;t<T;t++) th.emplace_back([&,t]{ for(size_t i=t;i<np;i+=T) memcpy(m+i*piece,src.data()+(i%64)*piece,piece);});
    for(auto&x:th)x.join(); munmap(m,N);
  } else if (mode==3){ // mmap parallel, contiguous big ranges per thread
    (void)!ftruncate(fd,N); char* m=(char*)mmap(0,N,PROT_READ|PROT_WRITE,MAP_SHARED|MAP_POPULATE,fd,0);
    std::vector<std::thread> th; size_t per=N/T;
    for(int t=0;t<T;t++) th.emplace_back([&,t]{ for(size_t o=0;o<per;o+=(1<<20)) memcpy(m+t*per+o,src.data(),1<<20);});
    for(auto&x:th)x.join(); munmap(m,N);
  } else if (mode==4){ // fallocate then parallel pwrite
    (void)!posix_fallocate(fd,0,N);
    double t1=now(); printf("fallocate %.3f s\n",t1-t0);
    std::vector<std::thread> th; size_t piece=1<<20; size_t np=N/piece;
    for(int t=0;t<T;t++) th.emplace_back([&,t]{ for(size_t i=t;i<np;i+=T) (void)!pwrite(fd,src.data()+(i%64)*piece,piece,i*piece);});
    for(auto&x:th)x.join();
  }
  double t1=now(); close(fd); unlink(path);
  printf("mode %d T %d: %.3f s  %.2f GB/s\n",mode,T,t1-t0,N/(t1-t0)/1e9);
}
